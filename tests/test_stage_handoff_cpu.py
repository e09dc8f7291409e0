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
