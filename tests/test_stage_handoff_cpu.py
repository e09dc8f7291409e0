"""Stage hand-off logic of the trainer (reference nnutils/train_utils.py:381-487) on the CPU: building the model and loading
a checkpoint need no kernel -- hypothesis selection, re-meshing to --n_faces, bone growth with k-means re-seeding."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimize                                              # noqa: E402
from lasr_amd.nnutils import mesh_net, train_utils           # noqa: E402


def opts_for(tmp, **over):
    flags = dict(name='t', checkpoint_dir=str(tmp), img_size=64, subdivide=2, n_bones=5, n_hypo=2, batch_size=1, num_epochs=1,
                 opt_tex='yes', use_gtpose=False, only_mean_sym=True, perceptual=False, use_graph=False)
    flags.update(over)
    argv = []
    for k, v in flags.items():
        argv += ['--%s%s' % ('' if v else 'no', k)] if isinstance(v, bool) else ['--%s' % k, str(v)]
    return optimize.parse_flags(argv)


def test_checkpoint_handoff_without_a_gpu(tmp_path):
    o = opts_for(tmp_path)
    tr = train_utils.LASRTrainer(o)
    tr.device = torch.device('cpu')
    torch.manual_seed(0)
    tr.model = mesh_net.LASR((64, 64), o, nz_feat=o.nz_feat)
    with torch.no_grad():
        tr.model.mean_v[1] *= torch.tensor([1.0, 0.7, 0.85])
        tr.model.ctl_ts.normal_()
    tr.epoch_nscore = torch.tensor([0.3, 0.1])                # hypothesis #1 is the better one
    tr.save('latest')
    ckpt = os.path.join(tr.save_dir, 'pred_net_latest.pth')
    old = tr.model

    # fewer hypotheses, symmetry dropped, same topology
    o2 = opts_for(tmp_path, name='t2', symmetric=False, n_hypo=1, n_faces='320', model_path=ckpt)
    net = mesh_net.LASR((64, 64), o2, nz_feat=o2.nz_feat)
    train_utils.LASRTrainer(o2).load_network(net, ckpt)
    assert torch.allclose(net.mean_v[0], old.symmetrize(old.mean_v[1]))
    assert torch.equal(net.code_predictor.quat_predictor.pred_layer.weight,
                       old.code_predictor.quat_predictor.pred_layer.weight.view(2, -1, o.nz_feat)[1])
    assert torch.equal(net.code_predictor.scale_predictor.pred_layer.bias, old.code_predictor.scale_predictor.pred_layer.bias[1:2])
    assert torch.equal(net.ctl_ts, old.ctl_ts.view(2, 4, 3)[1])
    assert torch.equal(net.encoder.enc_conv1[0].weight, old.encoder.enc_conv1[0].weight)

    # re-meshing (20 * 5^2 faces) and more bones: root bone rows kept, part bones from k-means on the new shape
    o3 = opts_for(tmp_path, name='t3', symmetric=False, n_hypo=1, n_bones=7, n_faces='500', model_path=ckpt)
    net3 = mesh_net.LASR((64, 64), o3, nz_feat=o3.nz_feat)
    fresh_q = net3.code_predictor.quat_predictor.pred_layer.weight.detach().clone()
    train_utils.LASRTrainer(o3).load_network(net3, ckpt)
    assert net3.faces.shape == (500, 3) and net3.mean_v.shape == (1, 252, 3) and float(net3.tex.detach().abs().max()) == 0
    q3 = net3.code_predictor.quat_predictor.pred_layer.weight.view(7, 4, -1)
    assert torch.equal(q3[0], old.code_predictor.quat_predictor.pred_layer.weight.view(2, 5, 4, -1)[1, 0])
    assert torch.equal(q3[1:], fresh_q.view(7, 4, -1)[1:])                      # new bones keep their fresh initialisation
    assert net3.rest_ts.shape == (6, 3) and torch.equal(net3.rest_ts, net3.ctl_ts)
    d = (net3.rest_ts[:, None] - net3.mean_v[0][None]).norm(dim=-1).min(1)[0]
    assert float(d.max()) < 0.5                                                  # centres lie in the (unit-scale) shape
    assert torch.equal(net3.log_ctl, torch.zeros_like(net3.log_ctl))             # a fresh model's values


# ---- exact-count, topology-preserving re-mesher (reference: Manifold + simplify, nnutils/train_utils.py:419-428) ---------------
def _point_triangle_distance(p, tri):
    """p [P,3], tri [F,3,3] -> [P] distance to the nearest triangle (Ericson, Real-Time Collision Detection 5.1.5)."""
    import numpy as np
    out = np.full(len(p), np.inf)
    cen = tri.mean(1)
    K = min(32, len(tri))                                 # exact test against the K triangles with the nearest centroids
    for s in range(0, len(p), 1024):
        q = p[s:s + 1024, None, :]
        near = np.argpartition(((q - cen[None]) ** 2).sum(-1), K - 1, axis=1)[:, :K]
        a, b, c = tri[near, 0], tri[near, 1], tri[near, 2]
        ab, ac = b - a, c - a
        ap = q - a
        d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
        bp = q - b
        d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
        cp = q - c
        d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
        vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
        den = np.where(np.abs(va + vb + vc) > 1e-300, va + vb + vc, 1.0)
        v, w = vb / den, vc / den
        cl = a + ab * v[..., None] + ac * w[..., None]                                   # interior
        def put(mask, pt):
            nonlocal cl
            cl = np.where(mask[..., None], pt, cl)
        t = np.clip(np.where(np.abs(d4 - d3 + d5 - d6) > 1e-300, (d4 - d3) / np.where(np.abs(d4 - d3 + d5 - d6) > 1e-300, d4 - d3 + d5 - d6, 1.0), 0.0), 0, 1)
        put((va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0), b + (c - b) * t[..., None])
        t = np.clip(np.where(np.abs(d2 - d6) > 1e-300, d2 / np.where(np.abs(d2 - d6) > 1e-300, d2 - d6, 1.0), 0.0), 0, 1)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + ac * t[..., None])
        t = np.clip(np.where(np.abs(d1 - d3) > 1e-300, d1 / np.where(np.abs(d1 - d3) > 1e-300, d1 - d3, 1.0), 0.0), 0, 1)
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + ab * t[..., None])
        put((d6 >= 0) & (d5 <= d6), np.broadcast_to(c, cl.shape))
        put((d3 >= 0) & (d4 <= d3), np.broadcast_to(b, cl.shape))
        put((d1 <= 0) & (d2 <= 0), np.broadcast_to(a, cl.shape))
        out[s:s + 1024] = np.linalg.norm(q - cl, axis=-1).min(1)
    return out


def _surface_samples(v, f):
    import numpy as np
    t = v[f]
    return np.concatenate([v, t.mean(1), (t[:, 0] + t[:, 1]) / 2, (t[:, 1] + t[:, 2]) / 2, (t[:, 2] + t[:, 0]) / 2])


def _two_lobed_shape():
    """A bent, waisted tube: two lobes joined by a neck, curved so that the centroid lies OUTSIDE the surface -- not star-shaped,
    the radial re-mesher of rounds 1-2 webbed such shapes over."""
    import numpy as np
    from lasr_amd import synth
    v, f = synth.geodesic_sphere(8)
    v = v.astype(np.float64)
    x = v[:, 0] * 1.6
    waist = 0.35 + 0.65 * np.abs(v[:, 0]) ** 1.5                       # thin neck in the middle, fat lobes at the ends
    y = v[:, 1] * 0.55 * waist + 1.1 * x ** 2                          # parabolic bend
    z = v[:, 2] * 0.55 * waist
    return np.stack([x, y, z], 1), f


def test_remesh_exact_counts_topology_and_hausdorff():
    import numpy as np
    from lasr_amd.nnutils import remesh
    v, f = _two_lobed_shape()
    tri = v[f]
    area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    centroid = (tri.mean(1) * area[:, None]).sum(0) / area.sum()
    t_last, t_first = remesh.ray_mesh_outermost(centroid, synth_dirs(), v, f, return_first=True)
    assert (np.isfinite(t_last) & (t_last - t_first > 0.05)).any()      # genuinely not star-shaped from its centroid
    diameter = float(np.linalg.norm(v[:, None] - v[None], axis=-1).max())
    for n in (1600, 1920, 2240, 2560, 2880):                             # scripts/template.sh:26-31, scripts/dog15.sh
        nv, nf = remesh.remesh_exact(v, f, n)
        assert nf.shape == (n, 3) and nv.dtype == np.float32 and nf.dtype == np.int64
        assert remesh._is_closed_manifold([tuple(t) for t in nf])       # closed, consistently oriented
        edges = {tuple(sorted(e)) for t in nf for e in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0]))}
        assert nv.shape[0] - len(edges) + n == 2                         # genus 0 kept
        assert nv.shape[0] == n // 2 + 2
        nt = nv[nf].astype(np.float64)
        # orientation kept: signed volume has the sign of the input's and nearly its value
        vol0 = (tri[:, 0] * np.cross(tri[:, 1], tri[:, 2])).sum() / 6
        vol1 = (nt[:, 0] * np.cross(nt[:, 1], nt[:, 2])).sum() / 6
        assert abs(vol1 - vol0) <= 0.02 * abs(vol0)
        if n not in (1600, 2880):
            continue
        # two-sided Hausdorff distance between the surfaces (dense samples of each against the other's triangles)
        d_new_to_old = _point_triangle_distance(_surface_samples(nv.astype(np.float64), nf), tri).max()
        d_old_to_new = _point_triangle_distance(_surface_samples(v, f), nt).max()
        assert max(d_new_to_old, d_old_to_new) < 0.02 * diameter, (n, d_new_to_old / diameter, d_old_to_new / diameter)
        # no slivers: every triangle keeps a sane aspect (area / longest edge^2)
        e = np.stack([np.linalg.norm(nt[:, i] - nt[:, (i + 1) % 3], axis=1) for i in range(3)], 1)
        a2 = np.linalg.norm(np.cross(nt[:, 1] - nt[:, 0], nt[:, 2] - nt[:, 0]), axis=1)
        assert (a2 / e.max(1) ** 2).min() > 0.02                      # the input's own thinnest triangle is at 0.034
    # decimation below the input's own count and the identity case
    nv, nf = remesh.remesh_exact(v, f, 500)
    assert nf.shape == (500, 3) and remesh._is_closed_manifold([tuple(t) for t in nf])
    nv, nf = remesh.remesh_exact(v, f, 1280)
    assert nf.shape == (1280, 3)
    import pytest
    with pytest.raises(ValueError):
        remesh.remesh_exact(v, f, 1601)                                  # odd face counts do not exist on closed meshes
    with pytest.raises(ValueError):
        remesh.remesh_exact(v, f[:-1], 1600)                             # an open mesh is refused, not silently patched


def synth_dirs():
    from lasr_amd import synth
    import numpy as np
    return synth.geodesic_sphere(4)[0].astype(np.float64)


def test_template_sh_stage_counts_survive_the_checkpoint_path(tmp_path):
    # scripts/template.sh stages 1-4 pass --n_faces 1600 / 1920 / 2240 / 2560 with --nosymmetric: each hand-off re-meshes the
    # previous stage's shape to exactly that count (rounds 1-2 snapped them to 1620 / 2000 / 2420 / 2420)
    o = opts_for(tmp_path, subdivide=3, n_hypo=1, n_bones=1)
    tr = train_utils.LASRTrainer(o)
    tr.device = torch.device('cpu')
    torch.manual_seed(0)
    tr.model = mesh_net.LASR((64, 64), o, nz_feat=o.nz_feat)
    tr.epoch_nscore = torch.zeros(1)
    tr.save('latest')
    ckpt = os.path.join(tr.save_dir, 'pred_net_latest.pth')
    for k, n in enumerate((1600, 1920, 2240, 2560)):
        ok = opts_for(tmp_path, name='s%d' % k, subdivide=3, symmetric=False, n_hypo=1, n_bones=1, n_faces=str(n), model_path=ckpt)
        net = mesh_net.LASR((64, 64), ok, nz_feat=ok.nz_feat)
        t2 = train_utils.LASRTrainer(ok)
        t2.load_network(net, ckpt)
        assert net.faces.shape == (n, 3) and net.mean_v.shape == (1, n // 2 + 2, 3)
        t2.device, t2.model, t2.epoch_nscore = torch.device('cpu'), net, torch.zeros(1)
        t2.save('latest')
        ckpt = os.path.join(t2.save_dir, 'pred_net_latest.pth')


def test_checkpoints_carry_the_reference_key_names_of_the_trunk(tmp_path):
    # the reference's predictor / extract tools look for encoder.resnet_conv.resnet.layerN.* (net_blocks.py:291-313)
    o = opts_for(tmp_path, name='k')
    tr = train_utils.LASRTrainer(o)
    tr.device = torch.device('cpu')
    torch.manual_seed(0)
    tr.model = mesh_net.LASR((64, 64), o, nz_feat=o.nz_feat)
    tr.save('latest')
    st = torch.load(os.path.join(tr.save_dir, 'pred_net_latest.pth'), map_location='cpu')
    own = tr.model.state_dict()
    trunk = [k for k in own if k.startswith('encoder.resnet_conv.')]
    assert trunk
    for k in trunk:
        ref = mesh_net.reference_resnet_key(k)
        assert ref is not None and ref.startswith('encoder.resnet_conv.resnet.') and torch.equal(st[ref], own[k]), k
        assert 'encoder.resnet_conv.' + mesh_net.map_resnet_key(ref) == k          # the two maps are inverse
    assert 'encoder.resnet_conv.resnet.layer1.0.conv1.weight' in st and 'encoder.resnet_conv.resnet.conv1.weight' in st
    net = mesh_net.LASR((64, 64), o, nz_feat=o.nz_feat)                            # and the file still loads here
    train_utils.LASRTrainer(o).load_network(net, os.path.join(tr.save_dir, 'pred_net_latest.pth'))
    assert torch.equal(net.encoder.resnet_conv.layers[1][0].conv1.weight, tr.model.encoder.resnet_conv.layers[1][0].conv1.weight)


def test_assigning_faces_drops_the_repeated_connectivity_caches(tmp_path):
    o = opts_for(tmp_path, name='f')
    net = mesh_net.LASR((64, 64), o, nz_feat=o.nz_feat)
    _, _, f1 = net.get_mean_shape(1)
    assert net._faces_n2 is not None
    new = net.faces.clone()
    new[0] = new[0].flip(0)
    net.faces = new                                                                # e.g. load_network / re-meshing
    assert net._faces_n2 is None and net._faces_rep is None
    _, _, f2 = net.get_mean_shape(1)
    assert torch.equal(f2[0], new) and not torch.equal(f1[0], f2[0])
    net.faces[1] = net.faces[1].flip(0)                                            # an in-place edit moves the version counter
    _, _, f3 = net.get_mean_shape(1)
    assert torch.equal(f3[0], net.faces)
