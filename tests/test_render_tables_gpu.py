"""fused_ops.render_tables (lasr_amd/csrc/post_raster.hip): one pass over the nine-attribute render and one pass back, against
  (1) the torch restatements of the reference lines it fuses (oracle/path_oracle.py: flow_reproject = nnutils/mesh_net.py:87-104,
      mask / flow / tex loss tables = :374-441, rndpair = :436-441), values and autograd gradients;
  (2) the separate HIP operators it replaces (image_losses.*, fused_ops.flow_reproject_planes) on contiguous copies of the
      planes: the tables must be BIT-IDENTICAL (same arithmetic, chunking and fold order)."""
import numpy as np
import pytest
import torch

from lasr_amd.nnutils import fused_ops, image_losses
from oracle import path_oracle as po

pytestmark = pytest.mark.gpu


def make_case(I, H, IS, seed=0, empty_flow_row=False):
    g = torch.Generator().manual_seed(seed)
    N, P = I * H, IS * IS
    px = torch.rand(N, 10, IS, IS, generator=g)
    px[:, 3:9] = px[:, 3:9] * 2 - 0.5
    px[:, 5] = px[:, 5] * 4 + 3                       # own depth
    px[:, 8] = px[:, 8] * 4 + 3                       # other frame's depth
    bgpix = torch.rand(N, IS, IS, generator=g) < 0.3  # background: the render leaves position 0 there
    px[:, 3:9] = torch.where(bgpix[:, None], torch.zeros(()), px[:, 3:9])
    masks = (torch.rand(I, IS, IS, generator=g) > 0.4).float()
    occ = torch.randn(I, IS, IS, generator=g)
    occ[occ.abs() < 0.3] = 0.0                        # occ == 0 marks unobserved pixels
    flow_obs = torch.randn(I, 3, IS, IS, generator=g) * 0.1
    imgs = torch.rand(I, 3, IS, IS, generator=g)
    pp = torch.randn(N, 2, generator=g) * 0.1
    fl = torch.rand(N, generator=g) + 8.0
    if empty_flow_row:
        masks[0] = 0.0                                # image 0 selects nothing for the flow term: NaN weights (kept behaviour)
    return px, masks, occ, flow_obs, imgs, pp, fl


def reference(px, masks, occ, flow_obs, imgs, pp, fl, H, wt, want_pair):
    """The fused lines as the reference writes them, on CPU torch."""
    N, I = px.shape[0], masks.shape[0]
    IS = px.shape[-1]
    half = N // 2
    other = torch.arange(N).roll(-half)
    rgb, pos6, alpha = px[:, :3], px[:, 3:9], px[:, 9]
    stacked = torch.cat([pos6, alpha[:, None]], 1)
    flow, bg = po.flow_reproject(stacked, pp, pp[other], fl[:, None], fl[other][:, None])
    fg = (masks > 0).float()[:, None]
    img_obs = imgs * fg
    img_white = 1 - fg + img_obs
    mask_tab = po.mask_loss_table(alpha.view(I, H, IS, IS), masks, occ)
    flow_tab, fmap = po.flow_loss_table(flow.view(I, H, IS, IS, 2), flow_obs, bg.view(I, H, IS, IS), occ, masks)
    tex_tab = po.tex_loss_table(img_obs, img_white, rgb.reshape(I, H, 3, IS, IS), alpha.view(I, H, IS, IS), occ, wt)
    pair = torch.cat([rgb * alpha[:, None], rgb], 0) if want_pair else None
    return mask_tab, flow_tab, tex_tab, flow, bg, fmap, pair


@pytest.mark.parametrize('I,H,IS,want_pair', [(2, 2, 24, True), (4, 1, 17, False), (2, 8, 64, True), (6, 1, 50, True)])
def test_values_and_gradients_against_the_reference_lines(cuda, I, H, IS, want_pair):
    px, masks, occ, flow_obs, imgs, pp, fl = make_case(I, H, IS, seed=I * 10 + H)
    wt = 0.7
    leaves = [t.clone().requires_grad_(True) for t in (px.double(), pp.double(), fl.double())]
    ref = reference(leaves[0], masks.double(), occ.double(), flow_obs.double(), imgs.double(), leaves[1], leaves[2], H, wt, want_pair)
    g = torch.Generator().manual_seed(5)
    cot = [torch.randn(I, H, generator=g).double() for _ in range(3)]
    cot_pair = torch.randn(2 * I * H, 3, IS, IS, generator=g).double() * 1e-3 if want_pair else None
    total = sum((c * t).sum() for c, t in zip(cot, ref[:3]))
    if want_pair:
        total = total + (cot_pair * ref[6]).sum()
    gref = torch.autograd.grad(total, leaves)

    d = lambda t: t.to(cuda)
    dpx, dpp, dfl = d(px).requires_grad_(True), d(pp).requires_grad_(True), d(fl).requires_grad_(True)
    obspair = fused_ops.obs_pair(d(imgs), d(masks))
    out = fused_ops.render_tables(dpx, d(masks), d(occ), d(flow_obs), obspair, dpp, dfl, wt, want_pair)
    for k, name in enumerate(('mask', 'flow', 'tex')):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), rtol=3e-5, atol=1e-7, err_msg=name)
    assert torch.equal(out[4].cpu(), ref[4]), 'bgmask'
    np.testing.assert_allclose(out[3].detach().cpu().numpy(), ref[3].detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out[5].cpu().numpy().reshape(I, H, IS, IS), ref[5].detach().numpy(), rtol=1e-4, atol=3e-6)   # |flow - obs| cancels in fp32
    sel = ~ref[4].view(I, H, IS, IS) & (occ != 0)[:, None] & (masks > 0)[:, None]
    assert torch.equal(out[6].cpu().view(I, H, IS, IS), sel), 'vis_mask'
    tot = sum((d(c).float() * t).sum() for c, t in zip(cot, out[:3]))
    if want_pair:
        np.testing.assert_allclose(out[7].detach().cpu().numpy(), ref[6].detach().numpy(), rtol=1e-6, atol=1e-7)
        tot = tot + (d(cot_pair).float() * out[7]).sum()
    gpx, gpp, gfl = torch.autograd.grad(tot, [dpx, dpp, dfl])
    for mine, theirs, name in ((gpx, gref[0], 'px'), (gpp, gref[1], 'pp'), (gfl, gref[2], 'fl')):
        scale = float(theirs.abs().max())
        assert float((mine.cpu().double() - theirs).abs().max()) <= 2e-4 * scale, name
    assert float(gpx[:, 3:6].abs().max()) == 0.0            # the rendering frame's own projection is detached (:101-102)


def test_tables_are_bit_identical_to_the_separate_operators(cuda):
    I, H, IS = 2, 8, 64
    px, masks, occ, flow_obs, imgs, pp, fl = [t.to(cuda) for t in make_case(I, H, IS, seed=3)]
    N = I * H
    other = torch.arange(N, device=cuda).roll(-(N // 2))
    obspair = fused_ops.obs_pair(imgs, masks)
    out = fused_ops.render_tables(px, masks, occ, flow_obs, obspair, pp, fl, 1.0, True)
    rgb, pos6, alpha = px[:, :3].contiguous(), px[:, 3:9], px[:, 9].contiguous()
    flow, bg = fused_ops.flow_reproject_planes(pos6, pp, pp[other], fl[:, None], fl[other][:, None])
    mask_tab = image_losses.mask_loss_table(alpha.view(I, H, IS, IS), masks, occ)
    flow_tab, fmap, vis = image_losses.flow_loss_table(flow.view(I, H, IS, IS, 2), flow_obs, bg.view(I, H, IS, IS), occ, masks, with_vis=True)
    tex_tab = image_losses.tex_loss_table(obspair[:I], obspair[I:], rgb.view(I, H, 3, IS, IS), alpha.view(I, H, IS, IS), occ, 1.0)
    assert torch.equal(out[0], mask_tab) and torch.equal(out[1], flow_tab) and torch.equal(out[2], tex_tab)
    assert torch.equal(out[3], flow) and torch.equal(out[4], bg)
    assert torch.equal(out[5].view_as(fmap), fmap) and torch.equal(out[6].view_as(vis), vis)
    assert torch.equal(out[7], torch.cat([rgb * alpha[:, None], rgb], 0))


@pytest.mark.parametrize('want_pair', [False, True])
def test_the_pass_from_the_images_forms_the_observed_pair_itself(cuda, want_pair):
    # render_tables_imgs = obs_pair + render_tables in one launch less: every output, the pair it hands back and the gradients are
    # the same bits (ragged size: chunks that do not divide the image)
    I, H, IS = 2, 3, 52
    px, masks, occ, flow_obs, imgs, pp, fl = [t.to(cuda) for t in make_case(I, H, IS, seed=5)]
    res = []
    for fused in (False, True):
        lp, lpp, lfl = (t.clone().requires_grad_(True) for t in (px, pp, fl))
        if fused:
            out = fused_ops.render_tables_imgs(lp, masks, occ, flow_obs, imgs, lpp, lfl, 0.7, want_pair)
            pair, out = out[-1], out[:-1]
        else:
            pair = fused_ops.obs_pair(imgs, masks)
            out = fused_ops.render_tables(lp, masks, occ, flow_obs, pair, lpp, lfl, 0.7, want_pair)
        tot = out[0].sum() * 1.3 + out[1].sum() * 0.4 + out[2].sum() * 2.1
        if want_pair:
            tot = tot + (out[7] * torch.linspace(-1, 1, out[7].numel(), device=cuda).view_as(out[7])).sum()
        tot.backward()
        res.append((out, pair, (lp.grad, lpp.grad, lfl.grad)))
    (oa, pa, ga), (ob, pb, gb) = res
    assert len(oa) == len(ob) == (8 if want_pair else 7) and torch.equal(pa, pb) and not pb.requires_grad
    assert all(torch.equal(a, b) for a, b in zip(oa, ob)) and all(torch.equal(a, b) for a, b in zip(ga, gb))


def test_an_image_without_flow_selection_gives_the_reference_nan_gradient(cuda):
    # mesh_net.py:408-412: the per-image weight normaliser is a mean over an empty selection -> NaN; the loss row is 0 but the
    # gradient is NaN, and the reference's trainer then drops the step (train_utils.py:289-290).  Kept on purpose.
    I, H, IS = 2, 1, 16
    px, masks, occ, flow_obs, imgs, pp, fl = [t.to(cuda) for t in make_case(I, H, IS, seed=9, empty_flow_row=True)]
    px.requires_grad_(True)
    out = fused_ops.render_tables(px, masks, occ, flow_obs, fused_ops.obs_pair(imgs, masks), pp, fl, 1.0, False)
    assert float(out[1][0, 0]) == 0.0 and torch.isfinite(out[1][1, 0])
    out[1].sum().backward()
    assert torch.isnan(px.grad[0, 6:9]).any() and torch.isfinite(px.grad[1]).all()


def test_cpu_tensors_are_rejected(cuda):
    px, masks, occ, flow_obs, imgs, pp, fl = make_case(2, 1, 8)
    with pytest.raises(TypeError):
        fused_ops.render_tables(px, masks, occ, flow_obs, torch.cat([imgs, imgs]), pp, fl)


# ---- the pre-raster side: fused_ops.raster_inputs ---------------------------------------------------------------------------
@pytest.mark.parametrize('n2,H,V', [(2, 8, 642), (4, 1, 1282), (6, 16, 37)])
def test_raster_inputs_against_the_reference_lines(cuda, n2, H, V):
    # nnutils/mesh_net.py:298-311 (pinhole of [x y z 1], near/far from the depth range), :81-82 / :354-355 (+ eye, y flip) and the
    # attribute triples of the three render calls (:85-87, :357): restated with torch ops + oracle/path_oracle.pinhole_cam
    g = torch.Generator().manual_seed(n2 + H + V)
    N = n2 * H
    cam = torch.randn(N, V, 3, generator=g)
    cam[:, :, 2] = cam[:, :, 2].abs() + 5
    tex = torch.rand(N, V, 3, generator=g)
    ppoint = torch.randn(n2, 2, generator=g) * 0.1
    scale = torch.rand(n2, H, generator=g) + 8
    eye = [0.0, 0.0, -2.732]
    up_pre, up_attr = torch.randn(N, V, 3, generator=g), torch.randn(N, V, 9, generator=g)

    leaves = [t.clone().double().requires_grad_(True) for t in (cam, tex, ppoint, scale)]
    c, t, pp, sc = leaves
    fl = po.pinhole_cam(torch.cat([c, torch.ones_like(c[:, :, :1])], -1), pp, sc)
    dmin, dmax = fl[:, :, 2].min(), fl[:, :, 2].max()
    near, far = dmin - (dmax - dmin) / 2, dmax + (dmax - dmin) / 2
    pre = (fl[:, :, :3] + torch.tensor(eye).double()[None, None]) * torch.tensor([1., -1., 1.]).double()
    other = c.reshape(2, N // 2, V, 3).flip(0).reshape(N, V, 3)
    attrs = torch.cat([t, c, other], -1)
    ((pre * up_pre.double()).sum() + (attrs * up_attr.double()).sum()).backward()

    d = [x.to(cuda).requires_grad_(True) for x in (cam, tex, ppoint, scale)]
    pp_all = d[2][:, None].repeat(1, H, 1).view(N, 2)
    got_pre, got_attrs, nf = fused_ops.raster_inputs(d[0], d[1], pp_all, d[3].reshape(N), eye)
    np.testing.assert_allclose(got_pre.detach().cpu().numpy(), pre.detach().numpy(), rtol=2e-6, atol=2e-6)
    assert torch.equal(got_attrs.detach().cpu(), attrs.detach().float())                       # copies: exact
    np.testing.assert_allclose(nf.cpu().numpy(), [float(near), float(far)], rtol=1e-6)
    ((got_pre * up_pre.to(cuda)).sum() + (got_attrs * up_attr.to(cuda)).sum()).backward()
    for mine, theirs, name in zip(d, leaves, ('verts_cam', 'tex', 'ppoint', 'scale')):
        scl = float(theirs.grad.abs().max())
        assert float((mine.grad.cpu().double() - theirs.grad).abs().max()) <= 2e-5 * scl, name
    # the pair of views the trainer stores in rasterizer.near / .far goes to the kernels without a launch
    from lasr_amd.soft_renderer.functional.soft_rasterize import _near_far_dev
    assert _near_far_dev(nf[0], nf[1], cuda).data_ptr() == nf.data_ptr()
