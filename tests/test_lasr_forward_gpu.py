"""End-to-end: LASR.forward / LASRTrainer.train_step on synthetic data, on the HIP kernels.
Checks the API contract of nnutils/mesh_net.py:152-556 (outputs, aux keys), that the fused loss tables agree with
the loop-based restatement on the very images the model rendered, that every parameter group receives a finite
gradient, and that optimisation reduces the loss."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import optimize                                                    # noqa: E402
from lasr_amd.nnutils import train_utils                            # noqa: E402
from oracle import path_oracle as po                                # noqa: E402

pytestmark = pytest.mark.gpu


def make_trainer(tmp_path, **over):
    flags = dict(name='t', checkpoint_dir=str(tmp_path), img_size=64, subdivide=2, n_bones=5, n_hypo=2, batch_size=2,
                 num_epochs=1, opt_tex='yes', use_gtpose=False, only_mean_sym=True, n_frames=3, iters_per_epoch=3,
                 perceptual=False, use_graph=False)        # eager unless a test asks for graph replay
    flags.update(over)
    argv = []
    for k, v in flags.items():
        if isinstance(v, bool):
            argv.append('--%s%s' % ('' if v else 'no', k))
        else:
            argv += ['--%s' % k, str(v)]
    opts = optimize.parse_flags(argv)
    torch.manual_seed(0)
    return train_utils.LASRTrainer(opts).init_training()


def test_forward_contract_and_loss_tables(tmp_path, cuda):
    tr = make_trainer(tmp_path)
    tr.model.train()
    tr.reinit_bones()
    m = tr.module
    batch = tr.set_input(tr.dataloader[0])
    loss, aux = tr.model(batch)
    assert loss.dim() == 0 and torch.isfinite(loss)
    for k in ('flow_rd_map', 'flow_rd', 'vis_mask', 'mask_pred', 'total_loss', 'mask_loss', 'texture_loss',
              'flow_rd_loss', 'triangle_loss', 'lmotion_loss', 'current_nscore', 'mask_hypo_0', 'tex_hypo_1',
              'texture_render', 'ctl_proj'):
        assert k in aux, k
    B, H, IS = 2, 2, 64
    assert aux['mask_pred'].shape == (2 * B * H, IS, IS) and aux['flow_rd'].shape == (2 * B * H, IS, IS, 2)
    assert aux['current_nscore'].shape == (H,)
    # silhouettes overlap the observation (the synthetic camera looks at the object)
    assert float(aux['mask_pred'].max()) > 0.9
    # fused tables == loop-based restatement on the same renders
    c = lambda t: t.detach().float().cpu()
    n2 = 2 * B
    ref = po.mask_loss_table(c(m.mask_pred).view(n2, H, IS, IS), c(m.masks), c(m.occ))
    np.testing.assert_allclose(c(m.mask_loss_sub).numpy(), ref.numpy(), rtol=1e-5, atol=1e-7)
    ref, _ = po.flow_loss_table(c(m.flow_rd).view(n2, H, IS, IS, 2), c(m.flow), c(m.bgmask).view(n2, H, IS, IS).bool(),
                                c(m.occ), c(m.masks))
    np.testing.assert_allclose(c(m.flow_rd_loss_sub).numpy(), ref.numpy(), rtol=1e-4, atol=1e-7)
    fg = (c(m.masks) > 0).float()[:, None]
    img_obs = c(m.imgs) * fg
    ref = 0.25 * po.tex_loss_table(img_obs, 1 - fg + img_obs, c(m.texture_render).view(n2, H, 3, IS, IS),
                                   c(m.mask_pred).view(n2, H, IS, IS), c(m.occ), 1.0)
    np.testing.assert_allclose(c(m.texture_loss_sub).numpy(), ref.numpy(), rtol=1e-4, atol=1e-7)


def test_every_parameter_group_gets_a_finite_gradient(tmp_path, cuda):
    tr = make_trainer(tmp_path)
    tr.model.train()
    tr.reinit_bones()
    loss, _ = tr.model(tr.set_input(tr.dataloader[0]))
    loss.backward()
    m = tr.module
    for name in ('mean_v', 'tex', 'ctl_rs', 'rest_ts', 'ctl_ts', 'log_ctl'):
        g = getattr(m, name).grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0, name
    enc = [p.grad for n, p in m.named_parameters() if n.startswith('encoder') and p.grad is not None]
    assert enc and all(torch.isfinite(g).all() for g in enc)
    assert any(float(g.abs().max()) > 0 for g in enc)


def test_training_reduces_the_loss_and_checkpoints(tmp_path, cuda):
    # Adam's first steps are sign-like, so run-to-run float noise (gather/scatter atomics in torch) moves individual
    # losses by several percent; compare window means over 40 iterations
    tr = make_trainer(tmp_path, iters_per_epoch=40, n_bones=1, n_hypo=1, batch_size=1)
    tr.model.train()
    losses = []
    for i, ids in enumerate(tr.dataloader):
        tr.module.iters = i
        l, _ = tr.train_step(tr.set_input(ids))
        losses.append(float(l))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < np.mean(losses[:10]), losses
    tr.epoch_nscore = torch.zeros(1, device=cuda)
    tr.save('latest')
    ckpt = os.path.join(tr.save_dir, 'pred_net_latest.pth')
    assert os.path.exists(ckpt)
    # stage hand-off: warm start a non-symmetric model of the same topology from the checkpoint (spot3.sh stage 0 -> 1)
    tr2 = make_trainer(tmp_path, name='t2', symmetric=False, n_bones=1, n_hypo=1, batch_size=1, n_faces='320', model_path=ckpt)
    full = tr.module.symmetrize(tr.module.mean_v[0]).detach()
    assert torch.allclose(tr2.module.mean_v[0], full, atol=1e-6)
    l, _ = tr2.model.train()(tr2.set_input(tr2.dataloader[0]))
    assert torch.isfinite(l)


def test_stage_handoff_selects_hypothesis_remeshes_and_grows_bones(tmp_path, cuda):
    # train_utils.py:381-487: fewer hypotheses -> the best one's predictor rows / shape / bones are kept; symmetry dropped
    # with a different --n_faces -> re-meshed; more bones -> root bone's rows kept, part bones re-seeded by k-means
    tr = make_trainer(tmp_path, n_bones=5, n_hypo=2)
    with torch.no_grad():
        tr.module.mean_v[1] *= torch.tensor([1.0, 0.7, 0.85], device=cuda)      # make the two hypotheses differ
    tr.epoch_nscore = torch.tensor([0.3, 0.1], device=cuda)                      # lower score = better: #1 wins
    tr.save('latest')
    ckpt = os.path.join(tr.save_dir, 'pred_net_latest.pth')
    tr2 = make_trainer(tmp_path, name='t2', symmetric=False, n_bones=5, n_hypo=1, batch_size=1, n_faces='320', model_path=ckpt)
    full = tr.module.symmetrize(tr.module.mean_v[1]).detach()
    assert torch.allclose(tr2.module.mean_v[0], full, atol=1e-6)
    old_q = tr.module.code_predictor.quat_predictor.pred_layer.weight.view(2, 5, 4, -1)[1]
    assert torch.allclose(tr2.module.code_predictor.quat_predictor.pred_layer.weight.view(5, 4, -1), old_q)
    assert torch.allclose(tr2.module.ctl_ts, tr.module.ctl_ts.view(2, 4, 3)[1])
    l, _ = tr2.model.train()(tr2.set_input(tr2.dataloader[0]))
    assert torch.isfinite(l)
    # re-meshing to --n_faces 500 (20 * 5^2 faces, 252 vertices) and 7 bones
    tr3 = make_trainer(tmp_path, name='t3', symmetric=False, n_bones=7, n_hypo=1, batch_size=1, n_faces='500', model_path=ckpt)
    m3 = tr3.module
    assert m3.faces.shape == (500, 3) and m3.mean_v.shape == (1, 252, 3) and m3.tex.shape == (1, 252, 3)
    r_old = (full - full.mean(0)).norm(dim=1)
    r_new = (m3.mean_v[0] - m3.mean_v[0].mean(0)).norm(dim=1)
    assert float(r_new.max()) <= float(r_old.max()) * 1.05 and float(r_new.min()) >= float(r_old.min()) * 0.85
    assert m3.rest_ts.shape == (6, 3) and float((m3.rest_ts - m3.ctl_ts).abs().max()) == 0
    d = (m3.rest_ts[:, None] - m3.mean_v[0][None]).norm(dim=-1).min(1)[0]
    assert float(d.max()) < 0.5 * float(r_old.max())                              # bone centres sit in the shape
    root = tr.module.code_predictor.depth_predictor.pred_layer.weight[0]
    assert torch.allclose(m3.code_predictor.depth_predictor.pred_layer.weight[0], root)      # root bone's rows kept
    tr3.model.train()
    for i in range(3):
        m3.iters = i
        l3, _ = tr3.train_step(tr3.set_input(tr3.dataloader[i]))
    assert torch.isfinite(l3)


def test_graph_replay_matches_eager_forward(tmp_path, cuda):
    # --use_graph replays forward+backward as one HIP graph; same batch + same parameters must give the eager loss
    tr = make_trainer(tmp_path, iters_per_epoch=6, use_graph=True)
    tr.model.train()
    tr.reinit_bones()
    m = tr.module
    for i in range(3):                                   # iteration 0 eager (part render), then capture + replay
        m.iters = i
        tr.train_step(tr.set_input(tr.dataloader[i]))
    assert hasattr(tr, '_graphs') and len(tr._graphs) == 1
    m.iters = 3
    batch = tr.set_input(tr.dataloader[3])
    with torch.no_grad():
        pass
    eager_loss, _ = tr.model({k: v.clone() for k, v in batch.items()})
    eager_loss = float(eager_loss.detach())
    graph_loss, _ = tr._graphed_forward_backward(batch, tr._graph_key())
    assert abs(float(graph_loss.detach()) - eager_loss) <= 1e-4 * max(1.0, abs(eager_loss))
    g = m.mean_v.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    torch.cuda.set_stream(torch.cuda.default_stream())   # the trainer switched the current stream; restore for other tests


@pytest.mark.parametrize('name,over', [
    # BASELINE configs[2]/[3]: camel stages 3-4 -- 2 pairs per GPU, one hypothesis, 36 bones, 512x512
    ('camel', dict(img_size=512, subdivide=3, n_bones=36, n_hypo=1, batch_size=2, symmetric=False, only_mean_sym=False)),
    # BASELINE configs[4]: dog15 at its real size (scripts/dog15.sh: 256x256, 3 pairs per GPU, 15 frames).
    # stage 0: 16 camera hypotheses, 21 bones (96 renders per call); stage 1: best hypothesis, 26 bones = the 25-bone LBS
    ('dog15-0', dict(img_size=256, subdivide=3, n_bones=21, n_hypo=16, batch_size=3, n_frames=15)),
    ('dog15-1', dict(img_size=256, subdivide=3, n_bones=26, n_hypo=1, batch_size=3, n_frames=15, symmetric=False,
                     only_mean_sym=False)),
])
def test_other_baseline_configurations_step(tmp_path, cuda, name, over):
    # trainer-level smoke test (train_step incl. the step tail) at these sizes; the PARITY of the forward / backward at the sizes
    # BASELINE names is tests/test_lasr_forward_oracle_gpu.py::test_whole_forward_*_at_size
    tr = make_trainer(tmp_path, name=name, iters_per_epoch=2, **over)
    tr.model.train()
    tr.reinit_bones()
    for i, ids in enumerate(tr.dataloader):
        tr.module.iters = i
        loss, aux = tr.train_step(tr.set_input(ids))
    assert torch.isfinite(loss)
    H = over['n_hypo']
    assert aux['current_nscore'].shape == (H,) and torch.isfinite(aux['current_nscore']).all()
    assert aux['mask_pred'].shape[0] == 2 * over['batch_size'] * H


def test_rendered_sequence_on_disk_trains(tmp_path, cuda):
    # SURVEY section 8 rows f4 + f2: scripts/render_syn.py writes a sequence in the reference's DAVIS layout, the video
    # loader reads it back, the trainer optimises on it (the path `optimize.py --dataname <seq>` takes)
    import importlib.util
    spec = importlib.util.spec_from_file_location('render_syn', os.path.join(ROOT, 'scripts', 'render_syn.py'))
    render_syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(render_syn)
    root = str(tmp_path / 'data')
    render_syn.main(['--outdir', 'syn-blob3f', '--nframes', '3', '--img_size', '128', '--root', root])
    seq = os.path.join(root, 'database', 'DAVIS')
    assert sorted(os.listdir(os.path.join(seq, 'JPEGImages', 'Full-Resolution', 'syn-blob3f'))) == \
        ['00000.jpg', '00001.jpg', '00002.jpg']
    assert os.path.exists(os.path.join(seq, 'FlowBW', 'Full-Resolution', 'syn-blob3f', 'flo-00002.pfm'))
    cam = np.loadtxt(os.path.join(seq, 'Camera', 'Full-Resolution', 'syn-blob3f', '00001.txt'))
    assert cam.shape == (8,) and cam[0] == 10 and cam[-1] == 10 and abs(np.linalg.norm(cam[3:7]) - 1) < 1e-6

    tr = make_trainer(tmp_path, dataname='syn-blob3f', data_root=root, batch_size=1)
    assert tr.sequence is None and tr.n_frames_on_disk == 3
    e = tr.dataloader.dataset[1]                                   # frames 0 -> 1
    fg = e['mask'][0] > 0
    assert 0.2 < fg.mean() < 0.7                                    # the object fills the 1.2x crop
    assert e['flow'][2][fg].mean() > 0.9 and np.isfinite(e['flow']).all()    # valid on the object (120 degree turns: large)
    tr.model.train()
    tr.reinit_bones()
    losses = []
    for i, batch in enumerate(tr.dataloader):
        tr.module.iters = i
        loss, aux = tr.train_step(tr.set_input(batch))
        losses.append(float(loss))
        if i == 5:
            break
    assert all(np.isfinite(losses))
    # extract.py: the checkpoint's per-frame shapes as .obj files, scored by eval_mesh.py against the rendered ground truth
    tr.epoch_nscore = torch.zeros(2, device=cuda)
    tr.save('latest')
    import extract
    flags = ['--name', 't', '--checkpoint_dir', str(tmp_path), '--img_size', '64', '--subdivide', '2', '--n_bones', '5', '--n_hypo', '2',
             '--batch_size', '1', '--opt_tex', 'yes', '--nouse_gtpose', '--only_mean_sym', '--noperceptual', '--dataname', 'syn-blob3f',
             '--data_root', root, '--model_path', os.path.join(tr.save_dir, 'pred_net_latest.pth')]
    out = extract.main(flags)
    assert sorted(out) == [0, 1, 2] and all(os.path.exists(p) for p in out.values())
    for fid in out:                                       # cam<i>.txt: [[R | T]; [fx fy ppx ppy]] (extract.py:123-130 of the reference)
        rtk = np.loadtxt(os.path.join(str(tmp_path), 't', 'cam%d.txt' % fid))
        assert rtk.shape == (4, 4) and np.isfinite(rtk).all()
        np.testing.assert_allclose(rtk[:3, :3] @ rtk[:3, :3].T, np.eye(3), atol=1e-4)      # a rotation
        assert rtk[2, 3] > 0 and rtk[3, 0] == rtk[3, 1] > 0                                # in front of the camera, fx == fy
    spec = importlib.util.spec_from_file_location('eval_mesh', os.path.join(ROOT, 'scripts', 'eval_mesh.py'))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    cds = ev.main(['--testdir', os.path.join(str(tmp_path), 't'), '--gtdir', os.path.join(seq, 'Meshes', 'Full-Resolution', 'syn-blob3f')])
    assert len(cds) == 3 and all(np.isfinite(cds)) and max(cds) < 10


def test_mesh_evaluation_protocol(cuda):
    # scripts/eval_mesh.py: a rigidly moved, rescaled copy scores ~0 (sampling noise only); a different shape does not
    import importlib.util
    from lasr_amd import synth
    spec = importlib.util.spec_from_file_location('eval_mesh', os.path.join(ROOT, 'scripts', 'eval_mesh.py'))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    v, f, _ = synth.blobby_mesh(8)
    v, f = torch.from_numpy(v).float().to(cuda), torch.from_numpy(np.asarray(f, np.int64)).to(cuda)
    a = 0.3
    R = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32, device=cuda)
    moved = (v @ R) * 3.0 + torch.tensor([0.5, -2.0, 1.0], device=cuda)
    same = ev.evaluate_pair((moved, f), (v, f))
    sphere = torch.nn.functional.normalize(v, dim=1)
    other = ev.evaluate_pair((sphere, f), (v, f))
    assert same < 0.05 and other > 5 * same


def test_graph_replay_covers_the_pose_noise_iterations(tmp_path, cuda):
    # epochs > 0 add random pose / scale noise on iterations 2..99 (mesh_net.py:220-235); its amplitude is a device scalar,
    # so these iterations replay one captured graph as well, drawing fresh random numbers on every replay
    tr = make_trainer(tmp_path, iters_per_epoch=8, use_graph=True, num_epochs=2)
    steps = tr.train()
    assert steps == 16 and {k[0] for k in tr._graphs} == {'plain', 'noisy'}
    idx = int(tr.module.optim_idx)
    m = tr.module
    m.epoch, m.iters = 1, 5
    batch = tr.set_input(tr.dataloader[0])
    with torch.no_grad():
        params = [p.detach().clone() for p in m.parameters()]
    losses = []
    for _ in range(2):                                  # same parameters, same batch: only the noise differs
        with torch.no_grad():
            for p, q in zip(m.parameters(), params):
                p.copy_(q)
        l, _ = tr.train_step(batch)
        losses.append(float(l))
    assert all(np.isfinite(losses)) and losses[0] != losses[1]
    # with two live graphs the parameters' .grad must follow the graph that was replayed last
    m.epoch, m.iters = 1, 150                                    # a plain iteration after noisy ones
    m.schedule_scalars()
    tr._graphed_forward_backward(batch, ('noisy', idx))
    tr._graphed_forward_backward(batch, ('plain', idx))
    g_graph = m.mean_v.grad.detach().clone()
    tr.optimizer.zero_grad(set_to_none=True)
    loss, _ = tr.model(batch)
    loss.mean().backward()
    g_eager = m.mean_v.grad.detach()
    # (eager and captured runs may pick different convolution algorithms: percent-level agreement; stale gradients of the
    # other graph would differ completely)
    assert float((g_graph - g_eager).abs().max()) <= 0.05 * float(g_eager.abs().max())


def test_render_syn_with_surface_textures(tmp_path, cuda):
    # scripts/render_syn.py --surface_tex = atlas image -> 5x5 per-face surface texels
    # (lasr_load_textures) -> hard rasteriser in surface-texture mode, the path the reference's render_syn.py:71 takes
    import importlib.util
    from PIL import Image
    spec = importlib.util.spec_from_file_location('render_syn', os.path.join(ROOT, 'scripts', 'render_syn.py'))
    render_syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(render_syn)
    imgs = {}
    for tag, extra in (('v', []), ('s', ['--surface_tex'])):
        root = str(tmp_path / tag)
        render_syn.main(['--outdir', 'syn', '--nframes', '2', '--img_size', '96', '--root', root] + extra)
        d = os.path.join(root, 'database', 'DAVIS')
        imgs[tag] = np.asarray(Image.open(os.path.join(d, 'JPEGImages', 'Full-Resolution', 'syn', '00000.jpg')), np.float32)
        mask = np.asarray(Image.open(os.path.join(d, 'Annotations', 'Full-Resolution', 'syn', '00000.png'))) > 0
        assert 0.05 < mask.mean() < 0.9
        assert os.path.exists(os.path.join(d, 'FlowFW', 'Full-Resolution', 'syn', 'flo-00000.pfm'))
    fg = imgs['s'][mask]
    assert fg.std(0).min() > 5                                  # the atlas pattern shows on the object
    assert np.abs(imgs['s'][mask] - imgs['v'][mask]).mean() > 5  # and differs from the vertex-coloured render


def test_normal_consistency_of_the_mesh_evaluation_on_analytic_meshes(cuda):
    # scripts/eval_mesh.py (reference :165-167,:197): the normal term of chamfer_distance on area-sampled points with face normals.
    # Two meshes whose normals are known in closed form: an octahedron (every face normal is (+-1, +-1, +-1) / sqrt(3)) and a
    # finely subdivided sphere (the normal of a face is the direction of its points up to the facet angle).
    import importlib.util
    from lasr_amd import synth
    spec = importlib.util.spec_from_file_location('eval_mesh', os.path.join(ROOT, 'scripts', 'eval_mesh.py'))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    gen = torch.Generator(device=cuda).manual_seed(3)
    ov = torch.tensor([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=torch.float32, device=cuda)
    of = torch.tensor([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], device=cuda)
    p, nrm = ev.sample_points(ov, of, 4000, gen, True)
    assert torch.allclose(nrm.abs(), torch.full_like(nrm, 3 ** -0.5), atol=1e-6)
    assert torch.allclose((p * nrm).sum(1), torch.full((4000,), 3 ** -0.5, device=cuda), atol=1e-5)      # outward, on the plane x.n = 1/sqrt(3)
    assert torch.allclose(p.abs().sum(1), torch.ones(4000, device=cuda), atol=1e-5)                     # |x| + |y| + |z| = 1
    v, f = synth.geodesic_sphere(8)                                              # 642 vertices, 1280 faces
    f = torch.from_numpy(np.asarray(f, np.int64)).to(cuda)
    sphere = torch.nn.functional.normalize(torch.from_numpy(np.asarray(v)).float().to(cuda), dim=1)
    q, nq = ev.sample_points(sphere, f, 4000, gen, True)
    assert float((torch.nn.functional.normalize(q, dim=1) * nq).sum(1).abs().min()) > 0.97              # facet angle of 1280 faces
    # the metric: a mesh against itself (other samples), against its copy with every face flipped (abs_cosine: still ~1),
    # against another shape (lower); the octahedron against itself is 1 up to sampling at the edges
    same_cd, same_nc = ev.evaluate_pair((sphere, f), (sphere, f), with_normals=True)
    flip_cd, flip_nc = ev.evaluate_pair((sphere, f.flip(1)), (sphere, f), with_normals=True)
    bv, bf, _ = synth.blobby_mesh(8)
    blob, bf = torch.from_numpy(bv).float().to(cuda), torch.from_numpy(np.asarray(bf, np.int64)).to(cuda)
    other_cd, other_nc = ev.evaluate_pair((blob, bf), (sphere, f), with_normals=True)
    assert same_nc > 0.97 and abs(flip_nc - same_nc) < 0.01 and other_nc < same_nc - 0.02 and other_cd > 5 * same_cd
    # closed form for the normal term: every sampled point of octant (+,+,+) matched to a point of the SAME face gives cos = 1
    x, nx = ev.sample_points(ov, of, 6000, gen, True)
    y, ny = ev.sample_points(ov, of, 6000, gen, True)
    cd, norm = ev.chamfer_with_normals(x, nx, y, ny)
    frac_cross = float(((nx * ny[torch.cdist(x, y).argmin(1)]).sum(1).abs() < 0.99).float().mean())     # neighbours across an edge: |cos| = 1/3
    assert norm <= 2 * (2. / 3.) * frac_cross + 0.02 and norm < 0.1
