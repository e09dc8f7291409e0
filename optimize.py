#!/usr/bin/env python3
"""Per-video optimisation driver with the reference's command line (/root/reference/optimize.py:33-56 and the
flag definitions of nnutils/mesh_net.py:54-73, nnutils/train_utils.py:58-68, dataloader/vid.py:34-35), so the
`scripts/*.sh` invocations work unchanged:

  python -m torch.distributed.launch --master_port P --nproc_per_node=N optimize.py --name=spot3-0 \
      --checkpoint_dir log/ --only_mean_sym --nouse_gtpose --subdivide 3 --n_bones 21 --n_hypo 8 \
      --num_epochs 5 --dataname spot3 --sil_path none --ngpu N --batch_size 1 --opt_tex yes

absl is not installed here, so its flag syntax (--flag value, --flag=value, --boolflag / --noboolflag) is parsed
by the few lines below.  Data is the synthetic sequence of lasr_amd/synth_data.py.
"""
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULTS = dict(
    local_rank=0, ngpu=1, sil_path='none', use_gtpose=True,                       # optimize.py:33-36
    noise=True, symmetric=True, symmetric_loss=True, nz_feat=200, texture=True, symmetric_texture=True,
    subdivide=3, symidx=0, n_bones=1, n_faces='1280', n_hypo=1, only_mean_sym=False, dataname='fashion',
    opt_tex='no', rscale=1.0, l1tex_wt=1.0, sigval=1e-4,                         # nnutils/mesh_net.py:54-73
    name='exp_name', num_epochs=1000, learning_rate=1e-4, batch_size=8, checkpoint_dir='./logdir',
    model_path='', save_epoch_freq=1,                                           # nnutils/train_utils.py:58-68
    img_size=256, n_data_workers=1,                                               # dataloader/vid.py:34-35
    # additions of this build (not in the reference): synthetic data shape and the perceptual term switch
    # use_graph: replay forward + backward as one HIP graph (on by default on a GPU; --nouse_graph = eager as the reference)
    # encoder_weights / alexnet_weights: optional local state_dicts (torchvision resnet18 / alexnet, an LPIPS 'alex' net, or a
    # reference LASR checkpoint) for the two networks the reference takes ImageNet-pretrained; '' = random init (no network here)
    # fused_tail: clipping + NaN guard + AdamW as three multi-tensor HIP launches (--nofused_tail = torch.optim.AdamW every step)
    n_frames=3, iters_per_epoch=200, perceptual=True, use_graph=True, data_root='.', encoder_weights='', alexnet_weights='',
    fused_tail=True,
    # deterministic: run-to-run reproducible optimisation (MIOpen immediate mode with deterministic kernels, torch's
    # deterministic algorithms; every kernel of lasr_amd is deterministic by construction).  Slower convolutions.
    deterministic=False,
    # overlap_allreduce: under torch.distributed with --use_graph, cut the captured backward at the encoder's layer-3 output and
    # all-reduce the gradients that exist by then while the rest of the backward replays (--nooverlap_allreduce: one message
    # after the whole replay)
    overlap_allreduce=True)


def parse_flags(argv, defaults=DEFAULTS):
    opts = dict(defaults)
    i = 0
    while i < len(argv):
        a = argv[i]
        i += 1
        if not a.startswith('--'):
            raise SystemExit('unexpected argument %r' % a)
        key, eq, val = a[2:].partition('=')
        key = key.replace('-', '_')
        if key not in opts and key.startswith('no') and isinstance(opts.get(key[2:]), bool):
            opts[key[2:]] = False
            continue
        if key not in opts:
            raise SystemExit('unknown flag --%s' % key)
        cur = opts[key]
        if isinstance(cur, bool):
            if eq:
                opts[key] = val.lower() in ('1', 'true', 'yes')
            elif i < len(argv) and argv[i].lower() in ('true', 'false'):
                opts[key] = argv[i].lower() == 'true'
                i += 1
            else:
                opts[key] = True
            continue
        if not eq:
            if i >= len(argv):
                raise SystemExit('flag --%s needs a value' % key)
            val = argv[i]
            i += 1
        opts[key] = type(cur)(val)
    return SimpleNamespace(**opts)


def main(argv):
    opts = parse_flags(argv)
    if 'LOCAL_RANK' in os.environ:                          # torch.distributed.run exports it instead of --local_rank
        opts.local_rank = int(os.environ['LOCAL_RANK'])
    from lasr_amd.nnutils import train_utils
    import torch.distributed as dist
    if torch.cuda.is_available():
        torch.cuda.set_device(opts.local_rank)
        torch.backends.cudnn.benchmark = not getattr(opts, 'deterministic', False)   # optimize.py:30-31 (on ROCm: MIOpen's find mode)
    world = int(os.environ.get('WORLD_SIZE', opts.ngpu))
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl' if torch.cuda.is_available() else 'gloo', init_method='env://')
    torch.manual_seed(0)
    torch.cuda.manual_seed(1) if torch.cuda.is_available() else None
    np.random.seed(0)
    random.seed(0)
    trainer = train_utils.LASRTrainer(opts)
    trainer.init_training()
    steps = trainer.train()
    if trainer.rank == 0:
        print('finished %d iterations' % steps)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1:])
