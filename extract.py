#!/usr/bin/env python3
"""Export the reconstructed shape of every frame of a sequence from a checkpoint (reference: /root/reference/extract.py,
which also renders visualisations with pyrender / matplotlib -- not reproduced).

    python extract.py --model_path log/spot3-1/pred_net_latest.pth --dataname spot3 --n_bones 26 --n_faces 1600 \
                      --nosymmetric --checkpoint_dir log/ --name spot3-1

For frame i of the sequence it writes <checkpoint_dir>/<name>/pred<i>.obj (articulated shape in camera space, the frame
the reference's scripts/eval_mesh.py evaluates: eval_mesh.py:106-109) and cam<i>.txt in the layout of the reference's
extract.py:123-130: np.savetxt of the 4x4 array [[R | T] (3x4, root body-to-camera transform); [fx, fy, ppx, ppy]] with
the intrinsics expressed in the uncropped image (nnutils/predictor.py:188-189).  The reference additionally writes bone /
skinning-Gaussian .ply files and novel-view renders for its visualisation scripts; those are not produced.  The flags are optimize.py's; the model is rebuilt as that stage built it and the checkpoint
is loaded as is (no re-meshing, no hypothesis selection beyond picking the best one for the export).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import optimize                                             # noqa: E402
from lasr_amd.nnutils import train_utils                    # noqa: E402
from lasr_amd.soft_renderer.functional import save_obj       # noqa: E402


def export(tr, out_dir):
    """-> {frame id: obj path}"""
    m = tr.module
    H = tr.opts.n_hypo
    score = getattr(tr, 'epoch_nscore', None)
    best = int((-score).argmax()) if (score is not None and H > 1) else 0
    os.makedirs(out_dir, exist_ok=True)
    n_frames = getattr(tr, 'n_frames_on_disk', None) or tr.opts.n_frames
    done = {}
    tr.model.train()                                         # the training-mode forward is the one that builds the geometry
    with torch.no_grad():
        for batch in tr.dataloader:
            m.iters = 1
            bi = tr.set_input(batch)
            tr.model(bi)
            ids = bi['frameid'].view(-1, 2).t().reshape(-1)  # undo the pair interleave
            verts = m.verts_cam.view(len(ids), H, -1, 3)[:, best]
            cam = {k: v.view(len(ids), H, *v.shape[1:])[:, best].cpu().numpy() for k, v in m.cam_export.items()}
            for k, fid in enumerate(int(v) for v in ids.tolist()):
                if fid in done:
                    continue
                path = os.path.join(out_dir, 'pred%d.obj' % fid)
                save_obj(path, verts[k].cpu(), m.faces.cpu())
                rtk = np.concatenate([np.concatenate([cam['R'][k], cam['T'][k][:, None]], 1),
                                      np.concatenate([cam['focal'][k], cam['pp'][k]])[None]], 0)
                np.savetxt(os.path.join(out_dir, 'cam%d.txt' % fid), rtk)
                done[fid] = path
            if len(done) >= n_frames:
                break
    return done


def main(argv):
    opts = optimize.parse_flags(argv)
    if not opts.model_path:
        raise SystemExit('--model_path is required')
    ckpt, opts.model_path = opts.model_path, ''              # build the model plainly, then load the tensors verbatim
    torch.manual_seed(0)
    tr = train_utils.LASRTrainer(opts).init_training()
    states = torch.load(ckpt, map_location='cpu')
    m = tr.module
    if 'faces' in states and states['faces'].shape != m.faces.shape:          # a re-meshed stage: adopt its topology
        m.faces = states['faces'].to(m.faces.device)
        m.mean_v.data = states['mean_v'].to(m.mean_v.device)
        m.tex.data = states['tex'].to(m.tex.device)
        tr.define_criterion_ddp()
    own = m.state_dict()
    m.load_state_dict({k: v for k, v in states.items() if k in own and torch.is_tensor(v) and own[k].shape == v.shape},
                      strict=False)
    if states.get('epoch_nscore') is not None and len(states['epoch_nscore']) == opts.n_hypo:
        tr.epoch_nscore = states['epoch_nscore'].to(tr.device)
    out = export(tr, os.path.join(opts.checkpoint_dir, opts.name))
    print('wrote %d meshes to %s' % (len(out), os.path.join(opts.checkpoint_dir, opts.name)))
    return out


if __name__ == '__main__':
    main(sys.argv[1:])
