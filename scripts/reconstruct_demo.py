#!/usr/bin/env python3
"""End-to-end check on data with known ground truth: render a synthetic turn-table sequence (scripts/render_syn.py), run the
two optimisation stages of scripts/spot3.sh on it through optimize.py's trainer, export the articulated shape of every
frame and score it against the ground-truth meshes with scripts/eval_mesh.py's protocol (diameter 10, rigid ICP, Chamfer).

    python scripts/reconstruct_demo.py [--nframes 3] [--epochs0 5] [--epochs1 10] [--out profiles/reconstruction.json]

Everything runs from random initialisation (no pretrained encoder offline).  Reports iterations/s of both stages as well.
"""
import argparse
import importlib.util
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimize                                             # noqa: E402
from lasr_amd.nnutils import train_utils                    # noqa: E402
from lasr_amd.soft_renderer.functional import load_obj      # noqa: E402


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'scripts', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_stage(argv):
    opts = optimize.parse_flags(argv)
    torch.manual_seed(0)
    np.random.seed(0)
    tr = train_utils.LASRTrainer(opts).init_training()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = tr.train()
    torch.cuda.synchronize()
    return tr, steps, time.perf_counter() - t0


def export_shapes(tr):
    """frame id -> articulated shape [V,3] in camera space of the selected hypothesis (extract.py), read back from the files."""
    import extract
    out_dir = os.path.join(tr.save_dir, 'export')
    paths = extract.export(tr, out_dir)
    dev = tr.device
    shapes = {i: load_obj(p, device=dev)[0].float() for i, p in paths.items()}
    return shapes, tr.module.faces.clone()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--nframes', type=int, default=3)
    ap.add_argument('--epochs0', type=int, default=5)
    ap.add_argument('--epochs1', type=int, default=10)
    ap.add_argument('--n_hypo', type=int, default=8)
    ap.add_argument('--out', default='')
    ap.add_argument('--no_graph', action='store_true')
    ap.add_argument('--deterministic', action='store_true', help='optimize.py --deterministic: the same command twice gives the same numbers')
    ap.add_argument('--img_size', type=int, default=256)
    ap.add_argument('--obj', default='', help='render this .obj instead of the built-in blobby sphere (render_syn.py --obj)')
    ap.add_argument('--model', default='', help="render_syn.py --model (placement preset, e.g. 'spot')")
    ap.add_argument('--surface_tex', action='store_true', help='render_syn.py --surface_tex')
    args = ap.parse_args(argv)
    root = tempfile.mkdtemp(prefix='lasr_demo_')
    name = 'syn-blob%df' % args.nframes
    extra = (['--obj', args.obj] if args.obj else []) + (['--model', args.model] if args.model else []) + \
        (['--surface_tex'] if args.surface_tex else [])
    _load('render_syn').main(['--outdir', name, '--nframes', str(args.nframes), '--root', root] + extra)
    ev = _load('eval_mesh')
    log = os.path.join(root, 'log')
    common = ['--checkpoint_dir', log, '--dataname', name, '--data_root', root, '--sil_path', 'none', '--ngpu', '1',
              '--batch_size', '1', '--opt_tex', 'yes', '--nouse_gtpose', '--subdivide', '3', '--img_size', str(args.img_size)] + \
        (['--nouse_graph'] if args.no_graph else ['--use_graph']) + (['--deterministic'] if args.deterministic else [])
    # scripts/spot3.sh:24-25
    tr0, steps0, dt0 = run_stage(['--name', 'demo-0', '--only_mean_sym', '--n_bones', '21', '--n_hypo', str(args.n_hypo),
                                  '--num_epochs', str(args.epochs0)] + common)
    dev = tr0.device
    gts = [load_obj(os.path.join(root, 'database', 'DAVIS', 'Meshes', 'Full-Resolution', name, '%05d.obj' % i), device=dev)
           for i in range(args.nframes)]

    def score(tr):
        shapes, faces = export_shapes(tr)
        return [ev.evaluate_pair((shapes[i].float(), faces.long()), (gts[i][0].float(), gts[i][1].long()))
                for i in sorted(shapes)]
    from lasr_amd import synth
    sv, sf = synth.geodesic_sphere(8)
    template = torch.from_numpy(sv).to(dev)
    cd_sphere = [ev.evaluate_pair((template, torch.from_numpy(sf).to(dev)), (g[0].float(), g[1].long())) for g in gts]
    cd0 = score(tr0)
    ckpt = os.path.join(log, 'demo-0', 'pred_net_latest.pth')
    del tr0
    torch.cuda.empty_cache()
    tr1, steps1, dt1 = run_stage(['--name', 'demo-1', '--nosymmetric', '--n_bones', '26', '--n_faces', '1600', '--n_hypo', '1',
                                  '--num_epochs', str(args.epochs1), '--model_path', ckpt] + common)
    cd1 = score(tr1)
    out = {'sequence': '%d frames, %s, full turn, 512x512 source' % (args.nframes, os.path.basename(args.obj) if args.obj else 'blobby sphere'),
           'chamfer_unit_sphere_template': float(np.mean(cd_sphere)),
           'stage0': {'iterations': steps0, 'seconds': dt0, 'iters_per_s': steps0 / dt0, 'chamfer': float(np.mean(cd0)), 'per_frame': cd0},
           'stage1': {'iterations': steps1, 'seconds': dt1, 'iters_per_s': steps1 / dt1, 'chamfer': float(np.mean(cd1)), 'per_frame': cd1},
           'protocol': 'scripts/eval_mesh.py: centred, diameter 10, rigid ICP, symmetric squared Chamfer of 10k samples'}
    print(json.dumps(out))
    if args.out:
        with open(args.out, 'w') as fh:
            json.dump(out, fh, indent=1)
    return out


if __name__ == '__main__':
    main()
